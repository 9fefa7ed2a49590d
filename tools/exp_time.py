import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, torch
import vslam_amd.capi as capi
lib = sys.argv[1]
capi.load_library.__defaults__ = (lib,)
from vslam_amd import synth
ctx = capi.Context(0)
bgr = synth.frames_torch(1, 128, 1280, 720, torch.device("cuda", 0))
gray = ctx.bgr2gray(bgr)
ctx.good_features(gray, 2000)
ctx.prof_enable(True)
for i in range(3):
    ctx.good_features(gray, 2000)
print(lib, {k: round(v[0] / v[1], 3) for k, v in ctx.prof_report().items()})
