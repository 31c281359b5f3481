"""Diagnostic: wall time of the BigVGAN vocoder on a configs[1]-length mel (1088 frames = 11.6 s of audio), bf16 and f32."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.vocoder import BigVGAN
dev = "cuda:0"
sd = W.synth_state_dict(W.vocoder_shapes(W.VOC_FULL), 0)
mel = (torch.randn(1, 100, 1088, generator=torch.Generator().manual_seed(1)) * 2 - 5).to(dev)
for dt in (sys.argv[1:] or ["bf16"]):
	v = BigVGAN(sd, W.VOC_FULL, dtype=dt, device=dev)
	v.inference(mel); torch.cuda.synchronize()
	ts = []
	for _ in range(3):
		t0 = time.perf_counter(); v.inference(mel); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
	print(f"BigVGAN {dt}: {min(ts):.2f} ms for 11.6 s of audio ({11.605 / (min(ts) * 1e-3):.0f}x real time)", flush=True)
	del v
