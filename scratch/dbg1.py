import sys, numpy as np, torch
sys.path.insert(0, '.')
import biggan_tensorflow_amd
from biggan_tensorflow_amd import functional as Fn
from oracle import ref_ops as R
from tests.common import rel_err, t2n
def cu(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    if grad: t.requires_grad_(True)
    return t
rng = np.random.default_rng(0)
for (M,K,N) in [(16,130,240),(16,34,64),(16,128,240),(16,130,128),(16,132,240)]:
    x, w, b = rng.standard_normal((M, K)), rng.standard_normal((K, N)) * 0.1, rng.standard_normal(N)
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    yr = xt @ wt; g = rng.standard_normal((M, N)); yr.backward(torch.tensor(g))
    z = np.zeros((M,256)); z[:, :K] = x
    zc = cu(z); wc = cu(w, True)
    y = Fn.DenseFn.apply(zc[:, :K], wc, None); y.backward(cu(g))
    print("dense", M,K,N, rel_err(t2n(y), yr.detach().numpy()), rel_err(t2n(wc.grad), wt.grad.numpy()))
for shape in [(130,240),(34,64),(128,240),(130,128)]:
    w = rng.standard_normal(shape) * 0.05
    wt = torch.tensor(w, requires_grad=True)
    Lr = R.ortho_reg_loss(wt, 1e-4, "ortho_cosine"); Lr.backward()
    wc = cu(w, True)
    L = Fn.OrthoCosineRegFn.apply(wc, 1e-4); L.backward()
    print("ortho", shape, abs(L.item()-Lr.item())/abs(Lr.item()), rel_err(t2n(wc.grad), wt.grad.numpy()))
