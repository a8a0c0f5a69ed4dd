import os, sys, collections, traceback, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench
from anyedit_amd.anysd.train import AnySDTrainer
dev = torch.device("cuda", 0)
unet, moe, sched = bench.build_model(dev)
for p in list(moe.image_proj_model.parameters()) + list(moe.adapter_modules) + [moe.task_embs]:
    p.requires_grad_(True)
B = 4
g = torch.Generator(device="cpu").manual_seed(4)
lat = torch.randn(B, 4, 64, 64, generator=g).to(dev); img = (torch.randn(B, 4, 64, 64, generator=g) * 0.18215).to(dev)
ehs = torch.randn(B, 77, 768, generator=g).to(dev); null = torch.randn(1, 77, 768, generator=g).to(dev)
ref = torch.randn(B, 257, 1280, generator=g).to(dev); code = (torch.arange(B) % 3).to(dev)
tr = AnySDTrainer(moe, sched.sqrt_alphas_cumprod, sched.sqrt_one_minus_alphas_cumprod, lr=1e-5)
def step(i):
    gi = torch.Generator(device="cpu").manual_seed(i)
    noise = torch.randn(B, 4, 64, 64, generator=gi).to(dev); t = torch.randint(0, 1000, (B,), generator=gi).to(dev); u = torch.rand(B, generator=gi).to(dev)
    return tr.train_step(lat, img, ehs, ref, code, noise, t, null_ehs=null.expand(B, -1, -1), dropout_u=u, dropout_p=0.05)
step(0); step(1); torch.cuda.synchronize()
counts = collections.Counter(); nbytes = collections.Counter()
orig_contig = torch.Tensor.contiguous; orig_to = torch.Tensor.to; orig_copy = torch.Tensor.copy_; orig_zeros = torch.zeros; orig_zl = torch.zeros_like; orig_cat = torch.cat
def site():
    for fr in traceback.extract_stack()[::-1]:
        if "anyedit_amd" in fr.filename and "prof_train_copies" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
def contig(self, *a, **k):
    if not self.is_contiguous():
        s = "contiguous " + site(); counts[s] += 1; nbytes[s] += self.numel() * self.element_size()
    return orig_contig(self, *a, **k)
def to(self, *a, **k):
    r = orig_to(self, *a, **k)
    if r is not self and self.is_cuda:
        s = "to " + site(); counts[s] += 1; nbytes[s] += self.numel() * self.element_size()
    return r
def zeros(*a, **k):
    r = orig_zeros(*a, **k); s = "zeros " + site(); counts[s] += 1; nbytes[s] += r.numel() * r.element_size(); return r
def zl(*a, **k):
    r = orig_zl(*a, **k); s = "zeros_like " + site(); counts[s] += 1; nbytes[s] += r.numel() * r.element_size(); return r
def cat(*a, **k):
    r = orig_cat(*a, **k); s = "cat " + site(); counts[s] += 1; nbytes[s] += r.numel() * r.element_size(); return r
torch.Tensor.contiguous = contig; torch.Tensor.to = to; torch.zeros = zeros; torch.zeros_like = zl; torch.cat = cat
step(2); torch.cuda.synchronize()
for s, b in nbytes.most_common(40):
    print(f"{b/1e6:9.1f} MB  x{counts[s]:4d}  {s}")
