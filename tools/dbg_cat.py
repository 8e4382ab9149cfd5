import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import OrderedDict
from garage_amd._lib import call, dptr, stream_ptr
from garage_amd.engine import pad_rows, reduction_workspace
from garage_amd.policies import CategoricalMLPPolicy
from garage_amd._dtypes import Box, Discrete, EnvSpec
from oracle import networks as nets
O, A, M = 5, 4, 8
spec = EnvSpec(Box(-np.inf, np.inf, (O,)), Discrete(A), max_episode_length=8)
torch.manual_seed(2)
pol = CategoricalMLPPolicy(spec, hidden_sizes=(16, 16), double_softmax=True)
dev = pol.device
rng = np.random.RandomState(3)
obs = torch.from_numpy(rng.randn(M, O).astype(np.float32))
act = torch.from_numpy(rng.randint(0, A, M).astype(np.float32))
adv = torch.ones(M)
sd = pol.state_dict()
print(list(sd.keys()))
with torch.no_grad():
    raw = nets.mlp_mean(sd, '_module.', obs)
    d = nets.categorical_dist(sd, '_module.', obs, True)
    ll = d.log_prob(act.long())
net = pol.net
X = pad_rows(obs)
scores = net.forward(X, M)
print('scores diff', (scores[:, :A].cpu() - raw).abs().max().item())
ll_out = torch.empty(M, device=dev); loss_out = torch.zeros(1, device=dev)
actd = pad_rows(act.reshape(-1, 1))
call('ga_ppo_categorical_loss_f32', dptr(scores), scores.stride(0), dptr(actd), actd.stride(0),
     None, dptr(adv.to(dev)), None, M, A, 1, 1, 0.2, 0.0, 0, None, dptr(ll_out), None,
     dptr(loss_out), None, None, 0, 0, dptr(reduction_workspace(dev)), stream_ptr())
print('ll gpu', ll_out.cpu().numpy())
print('ll cpu', ll.numpy())
print('loss', loss_out.item(), -(ll*adv).mean().item())
