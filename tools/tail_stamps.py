#!/usr/bin/env python3
"""Cycle stamps of the fused decoder tail (csrc/tail.hip built with -DWSI_STUDY into wsi_segmentation_pipeline_amd/lib_study/):
where do a conv1 wave and a conv2 wave of unet_tail2_kernel spend an interval?
build: cd wsi_segmentation_pipeline_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DWSI_STUDY -c tail.hip -o /tmp/tail_study.o
       && hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_study/libwsi_hip.so $(ls build/*.o | grep -v tail.o) /tmp/tail_study.o"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wsi_segmentation_pipeline_amd import native
native.LIB_PATH = os.path.join(ROOT, 'wsi_segmentation_pipeline_amd', 'lib_study', 'libwsi_hip.so')
from wsi_segmentation_pipeline_amd import synthetic as W
from wsi_segmentation_pipeline_amd.engine import PARITY
from wsi_segmentation_pipeline_amd.unet import UNetEngine

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
usd = W.make_unet_state_dict(5, classes=4)
eng = UNetEngine(usd, dev, planes=PARITY, max_batch=n)
lib = native.load()
lib.wsi_study_tail_stamps.argtypes = [C.c_void_p, C.c_int]
if len(sys.argv) > 2:
    lib.wsi_conv_set_mode(1 + int(sys.argv[2]))
side = 1
while side * side < n:
    side += 1
level = torch.randint(0, 256, (side * 256, side * 256, 3), dtype=torch.uint8, device=dev)
xy = torch.tensor([[256 * (i % side), 256 * (i // side)] for i in range(n)], dtype=torch.int32, device=dev)
eng.forward_tiles(level, xy, 256, 256)
torch.cuda.synchronize()
lib.wsi_study_tail_stamps(None, 1)
eng.forward_tiles(level, xy, 256, 256)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib.wsi_study_tail_stamps(buf, 0)
v = list(buf)
rep = max(1, v[0])
names = {1: 'conv1 wave: MFMA loops (2 tiles x 6 taps)', 2: 'conv1 wave: epilogue + LDS writes', 3: 'conv1 wave: poll of the read counter', 4: 'conv1 wave: DMA wait + barrier',
         5: 'conv2 wave: MFMA loop (2 tiles x 12 taps)', 6: 'conv2 wave: epilogue + head + stores', 7: 'conv2 wave: barrier', 8: 'kernel (workgroup lifetime)'}
print('%d reporting workgroups; shader cycles (s_memtime) per workgroup, set_mode flags %s' % (rep, sys.argv[2] if len(sys.argv) > 2 else '0'))
for i in range(1, 9):
    print('  %-48s %10.1f' % (names[i], v[i] / rep))
