"""Inputs of tools/pcl_golden.cpp: the synthetic frames the tests use (perception_amd.synth, seed 20190409 + i) as raw
little-endian float32 records x y z rgb, and the launch-default template as an ASCII PCD.
usage: python tools/write_synth_frames.py <dir> [n_frames=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perception_amd import pcd, synth, templates
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.makedirs(d, exist_ok=True)
for i in range(n):
    synth.frame(i).astype("<f4").tofile(os.path.join(d, "frame_%d.bin" % i))
pcd.write_pcd_ascii(os.path.join(d, "template.pcd"), templates.make_cuboid_template(**templates.DEFAULT_TEMPLATE))
print("wrote %d frames and template.pcd to %s" % (n, d))
