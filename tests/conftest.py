import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
	if p not in sys.path:
		sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
	# the CPU oracle is most of the GPU suite's wall time.  A 1-GPU box gives the job a 16-core share whatever os.cpu_count() says, and torch sizes its
	# intra-op pool by cpu_count: oversubscribed, the oracle's GEMMs crawl (bench.py's cpu_baseline caps its threads for the same reason)
	try:
		import torch
		try:
			cores = len(os.sched_getaffinity(0))
		except AttributeError:
			cores = os.cpu_count() or 1
		torch.set_num_threads(max(1, min(cores, 16)))
	except ImportError:
		pass


@pytest.fixture(scope="session")
def golden():
	import numpy as np

	def load(name):
		return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
	return load
