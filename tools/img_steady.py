"""Image branch only: warm-up (MIOpen find mode), a marker kernel, then 10 steady-state runs (for rocprofv3 traces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.model import default_args
from dagr_amd.model.networks.dagr import DAGR
from dagr_amd.utils.testing_weights import randomize_
torch.backends.cudnn.benchmark = True
a = default_args(batch_size=8, use_image=True, img_net="resnet50")
m = randomize_(DAGR(a, height=480, width=640)).eval().cuda()
eng = m.engine()
img = torch.rand(8, 3, 480, 640, device="cuda")
with torch.no_grad():
    for _ in range(3):
        eng._image_branch(img)
    torch.cuda.synchronize()
    marker = torch.zeros(7, device="cuda").cumsum(0)      # a kernel that appears nowhere else: start of the steady part
    torch.cuda.synchronize()
    for _ in range(10):
        eng._image_branch(img)
    torch.cuda.synchronize()
