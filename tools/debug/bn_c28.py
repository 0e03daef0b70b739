import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from abi_call import Abi
from action_conditioned_gans_amd import _lib
abi = Abi(_lib.get(), 'cuda:0')
torch.manual_seed(0)
for rows in (131072, 32768, 8192):
    for c in (16, 24, 28, 32, 36, 60, 64):
        x = (torch.randn(rows, c) * 1.5 + 0.7)
        dy = torch.randn(rows, c)
        beta = torch.randn(c) * 0.3
        xd = x.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
        m = xd.mean(0); v = xd.var(0, unbiased=False)
        y_ref = torch.relu((xd - m) / torch.sqrt(v + 1e-3) + bd)
        dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
        xg = x.cuda().view(rows, 1, 1, c); dyg = dy.cuda().view(rows, 1, 1, c)
        y, mean, rstd = abi.bn_act_fwd(xg, beta.cuda(), 'relu', 1)
        dx, dbeta = abi.bn_act_bwd(xg, dyg, beta.cuda(), mean, rstd, 'relu', 1)
        ey = (y.cpu().view(rows, c).double() - y_ref).abs().amax(0)
        ed = (dx.cpu().view(rows, c).double() - dx_ref).abs().amax(0)
        eb = (dbeta.cpu().double() - db_ref).abs()
        print(rows, c, 'y err max %.2e' % ey.max(), 'dx err max %.2e' % ed.max(), 'bad dx channels', (ed > 1e-3).nonzero().flatten().tolist(), 'dbeta err %.2e' % eb.max(), flush=True)
