#!/usr/bin/env python3
"""Register / scratch use of every kernel of one csrc/*.hip file (device-only assembly, no GPU needed):
   python tools/kernel_regs.py conv_pp [filter]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'wsi_segmentation_pipeline_amd', 'csrc', sys.argv[1] + '.hip')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
extra = ['-DWSI_STUDY'] if os.environ.get('STUDY') else []
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, 'k.s')
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only', src, '-o', out] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
names = re.findall(r'\.amdhsa_kernel (\S+)', s)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
for (m, d) in zip(re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S), dem):
    body = m.group(2)
    if flt and flt not in d:
        continue
    nv = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)
    ao = re.search(r'\.amdhsa_accum_offset (\d+)', body)
    sp = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)
    print(d[:100].ljust(100), 'vgpr+agpr', nv, 'accum_offset', ao.group(1) if ao else '-', 'scratch', sp)
