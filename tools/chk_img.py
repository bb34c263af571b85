import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.model import default_args
from dagr_amd.model.networks.dagr import DAGR
from dagr_amd.utils.testing_weights import randomize_
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
a = default_args(batch_size=8, use_image=True, img_net="resnet50")
m = randomize_(DAGR(a, height=480, width=640)).eval().cuda()
eng = m.engine()
img = torch.rand(8, 3, 480, 640, device="cuda")
with torch.no_grad():
    f_ref, o_ref = m.backbone.net(img.contiguous(memory_format=torch.channels_last))
    feats, cnn = eng._image_branch(img)
    for a_, b_ in zip(feats, f_ref):
        print("feat max|diff| %.2e  max|ref| %.2e" % ((a_ - b_).abs().max().item(), b_.abs().max().item()))
    def t(f, n=10):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
    print("folded+fused image branch ms", t(lambda: eng._image_branch(img)))
