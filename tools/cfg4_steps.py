#!/usr/bin/env python3
"""Per-step wall times of the cfg4 workload (bench.py's construction), synchronised after every step and free-running."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import bags as B, synthetic as W
from wsi_segmentation_pipeline_amd.engine import TrunkEngine, MX
dev = torch.device('cuda:0')
sd = W.make_resnet18_state_dict(11, with_fc=True)
eng = TrunkEngine(sd, dev, planes=MX, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=6200 * 16)
wl = B.BagWorkload(eng, sd, 4000, seed=4, device=dev, rank=0, world=1)
ts = []
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wl.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('synchronised steps (ms):', ' '.join('%.1f' % t for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5):
    out = wl.step()
torch.cuda.synchronize()
print('5 free-running steps: %.1f ms per step' % ((time.perf_counter() - t0) * 1e3 / 5))
t0 = time.perf_counter()
for i in range(5):
    out = wl.step()
    del out
torch.cuda.synchronize()
print('5 free-running steps, result dropped each step: %.1f ms per step' % ((time.perf_counter() - t0) * 1e3 / 5))
